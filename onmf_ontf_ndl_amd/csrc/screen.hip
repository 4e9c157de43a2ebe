// Device-side screen of the sweep loop's stopping test (src/onmf.py:252, :265 of the reference):
//     continue while  dist = sigma_max(H_new - H_old) / sigma_max(H_old)  >  stopping_diff.
// Evaluating dist exactly needs two r x r Gram matrices and their top eigenvalues per sweep, as
// much arithmetic as the sweep itself.  The sweep kernel instead produces, as by-products,
//     lb2  = max_p || u_p^T (H_new - H_old) ||^2     for 16 unit probe vectors u_p   (<= sigma_max^2)
//     cw   = max_k y_k / v_k,  y = H (H^T v), v > 0   (Collatz-Wielandt: >= sigma_max(H)^2, because
//                                                      H H^T is entrywise nonnegative)
// and screen_verdict() turns them into a three-way verdict per sweep, evaluated by the NEXT sweep's
// workgroups in their prologue (the last sweep's by a small kernel of this file):
//     lower bound of dist > tol  -> keep sweeping (certain)
//     upper bound of dist <= tol -> stop (certain)
//     otherwise                  -> inconclusive: the host evaluates the test in full
// Bounds are rigorous for ANY probes, so the verdicts never differ from the full test; good probes
// only make "inconclusive" rare.  The probes follow the dominant left singular direction of
// H_new - H_old by one step of subspace iteration per sweep (Z accumulated by the sweep kernel),
// the top Ritz direction placed in probe 0 and the other probes kept orthogonal to it.
#include "common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int blk_off(int i, int j) { return i * 16 + 4 * ((j >> 2) ^ swz(i)) + (j & 3); }

__device__ __forceinline__ float hash_unit(unsigned x) {      // deterministic value in [-1, 1)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

// Workgroup 0 sets the screen's state up; the others (if any) draw the solve's start code, fill_n uniform
// floats keyed by fill_seed, exactly as fill_uniform_kernel would.
__global__ __launch_bounds__(256) void screen_init_kernel(float* base, int r, int keep_probes, int n_sets,
                                                          float* __restrict__ fill_out, size_t fill_n,
                                                          uint64_t fill_seed) {
    if (blockIdx.x > 0) {
        const size_t nquads = (fill_n + 3) / 4;
        for (size_t q = (blockIdx.x - 1) * (size_t)blockDim.x + threadIdx.x; q < nquads;
             q += (size_t)(gridDim.x - 1) * blockDim.x)
            fill_uniform_quad(fill_out, fill_n, q, fill_seed);
        return;
    }
    const int tid = threadIdx.x;
    const float scale = 1.0f / sqrtf((float)r);              // |u_p|^2 <= r * scale^2 = 1
    // keep_probes: the probe vectors left by the previous solve (same r) are a better start than
    // pseudo-random ones - consecutive minibatches of a stream move the code along similar directions
    const int first = keep_probes ? ScreenLayout::ZACC : 0;
    // (only the accumulator sets of the sweeps this solve can run)
    const int sets_end = ScreenLayout::SETS + n_sets * ScreenLayout::SETSZ;
    for (int e = first + tid; e < sets_end; e += 256) base[e] = 0.f;
    for (int e = ScreenLayout::CNT + tid; e < ScreenLayout::TOTAL; e += 256) base[e] = 0.f;
    __syncthreads();
    if (tid <= SCREEN_MAX_SWEEPS) base[ScreenLayout::UBH + tid] = -1.0f;     // no bound known
    if (tid < r && !keep_probes) {
        const int K = tid >> 4, j = tid & 15;
        for (int p = 0; p < 16; ++p)
            base[ScreenLayout::UBLK + K * 256 + blk_off(p, j)] = scale * hash_unit(tid * 16 + p + 1);
    }
    if (tid == 0) base[ScreenLayout::AVEC] = 1.0f;
    if (!keep_probes) base[ScreenLayout::CWV + tid] = 1.0f;
}

// Reopen the loop at sweep k after the host settled sweep k - 1.  The launch that was stopped had
// already run sweep k ahead of that verdict: its sums and its count of finished workgroups are
// stale and sweep k is about to add its own again, so the accumulator sets from k on start over.
__global__ void screen_clear_flag_kernel(float* base, int k, float ub) {
    for (int j = k; j < SCREEN_MAX_SWEEPS && j < k + 2; ++j) {
        for (int e = threadIdx.x; e < ScreenLayout::SETSZ; e += blockDim.x)
            base[ScreenLayout::SETS + j * ScreenLayout::SETSZ + e] = 0.f;
        if (threadIdx.x == 0) reinterpret_cast<int*>(base)[ScreenLayout::CNT + j] = 0;
    }
    if (threadIdx.x == 0) {
        reinterpret_cast<int*>(base)[ScreenLayout::FLAG] = 0;
        base[ScreenLayout::UBH + k] = ub;
        // the host settled sweep k - 1 as "continue": the device's own reading of the records
        // (screen_finalize_kernel) must see it that way too
        if (k > 0) base[ScreenLayout::REC + 4 * (k - 1) + 2] = 1.0f;
    }
}

// Power step of the probe subspace on a sample of the columns (16 of every `stride` columns):
//     Z += dH_S (U^T dH_S)^T,   dH = H_new - H_old
// One workgroup of 4 waves per sampled 16-column tile; wave w owns the row blocks J = w, w+4, ...
// Registers are in the MFMA layout of the sweep kernel (lane l: column l & 15, rows
// 16J + 4(l >> 4) + t).  P = U^T dH_S contracts over rows (native layout; the four waves' partial
// P are summed through LDS); the second product contracts over the 16 columns, so dH blocks and P
// pass through 16x17 LDS tiles to put the column index on the MFMA's k lanes.  Any U yields a
// valid bound, so a sample suffices.
template <int NB>
__global__ __launch_bounds__(256) void screen_refresh_kernel(int r, int b, int stride,
                                                             const float* __restrict__ Hnew,
                                                             const float* __restrict__ Hold, float* base) {
    constexpr int NJ = (NB + 3) / 4;          // row blocks per wave
    __shared__ float sd[4][16 * 17];
    __shared__ float sp[16 * 17];
    if (reinterpret_cast<const int*>(base)[ScreenLayout::FLAG] != 0) return;
    const float* ublk = base + ScreenLayout::UBLK;
    float* zacc = base + ScreenLayout::ZACC + (blockIdx.x % SCREEN_ZREPL) * 4096;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int col = blockIdx.x * stride + c;
    const bool cvalid = col < b;
    const int sw_c = swz(c);
    const unsigned nbytes = (unsigned)r * (unsigned)b * 4u;
    const auto rs_n = __builtin_amdgcn_make_buffer_rsrc((void*)Hnew, 0, nbytes, 0x00020000);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)Hold, 0, nbytes, 0x00020000);
    const int voff = (4 * g * b + min(col, b - 1)) * 4;
    for (int e = tid; e < 16 * 17; e += 256) sp[e] = 0.f;
    float dh[NJ][4];
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int J = wave + 4 * jj;           // rows past r read as 0 through the buffer range check
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float hn = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_n, voff, (16 * J + t) * b * 4, 0));
            const float ho = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_o, voff, (16 * J + t) * b * 4, 0));
            dh[jj][t] = (cvalid && J < NB) ? (hn - ho) : 0.f;
        }
        const int Jc = min(J, NB - 1);
        const float4 u4 = *reinterpret_cast<const float4*>(ublk + Jc * 256 + c * 16 + 4 * (g ^ sw_c));
        pa = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.x, dh[jj][0], pa, 0, 0, 0);
        pb2 = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.y, dh[jj][1], pb2, 0, 0, 0);
        pa = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.z, dh[jj][2], pa, 0, 0, 0);
        pb2 = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.w, dh[jj][3], pb2, 0, 0, 0);
    }
    __syncthreads();
    // partial P[probe 4g+t][column c] of this wave -> sum over the four waves
#pragma unroll
    for (int t = 0; t < 4; ++t) atomicAdd(&sp[(4 * g + t) * 17 + c], pa[t] + pb2[t]);
    __syncthreads();
    float pb[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) pb[s4] = sp[c * 17 + 4 * s4 + g];       // B[k = col 4s+g][j = probe c]
    float* sdw = sd[wave];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int K = wave + 4 * jj;
        if (K < NB) {                          // wave-uniform
#pragma unroll
            for (int t = 0; t < 4; ++t) sdw[(4 * g + t) * 17 + c] = dh[jj][t];
            __builtin_amdgcn_wave_barrier();
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)                                 // A[i = row c][k = col 4s+g]
                z = __builtin_amdgcn_mfma_f32_16x16x4f32(sdw[c * 17 + 4 * s4 + g], pb[s4], z, 0, 0, 0);
            __builtin_amdgcn_wave_barrier();
            // z[t] = Z[row 16K+4g+t][probe c]
#pragma unroll
            for (int t = 0; t < 4; ++t) atomicAdd(&zacc[(16 * K + 4 * g + t) * 16 + c], z[t]);
        }
    }
}

template <int NB>
int run_refresh(onmf_ctx* c, int r, int b, const float* Hnew, const float* Hold, float* base,
                hipStream_t s) {
    const int stride = 256;
    const int tiles = (b + stride - 1) / stride;
    hipLaunchKernelGGL((screen_refresh_kernel<NB>), dim3(tiles), dim3(256), 0, s, r, b, stride, Hnew, Hold, base);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

// The verdicts that no sweep evaluated (a sweep evaluates the one two sweeps back): sweeps
// k_from .. k_last in order, up to the first that ends the loop (one wave).
// `rec_host` (may be null): host memory mapped into the device's address space; the verdict records and the
// words behind them are copied there at the end, so that the host can read them once this kernel is done -
// no copy kernel (4 us and a 6 us gap behind it) between the verdicts and the work that follows them.
__global__ __launch_bounds__(64) void screen_final_kernel(float* base, int k_from, int k_last, float tol,
                                                          float rank_cap, const int* start_dev, int cw_rows,
                                                          float* __restrict__ rec_host) {
    bool open = reinterpret_cast<const int*>(base)[ScreenLayout::FLAG] == 0;   // else the loop ended earlier (uniform)
    if (open && start_dev != nullptr) {
        // behind a launch whose first sweep came from device memory: that launch evaluated the tests from
        // its first sweep up to two before its last
        const int first = *start_dev;
        if (first > k_last) open = false;
        else k_from = max(first, k_last - 1);
    }
    if (open) {
        float ub = base[ScreenLayout::UBH + k_from];
        for (int j = k_from; j <= k_last; ++j)
            if (screen_eval_wave(base, j, ub, tol, rank_cap, true, -1.0f, cw_rows) != 0) break;
    }
    if (cw_rows > 0) {
        // One power step of the Collatz-Wielandt vector for the next solve, on the sums of the last sweep
        // that ran (its count of workgroups is positive): v <- y / max y, kept strictly positive (rows of
        // H that are all zero would zero it).  Whatever v is, the bound it gives is valid - as long as the
        // sums were formed with the same v: so only now, after this solve's last verdict.
        const int lane = threadIdx.x;
        int last = -1;
        for (int j = k_last; j >= 0 && last < 0; --j)
            if (reinterpret_cast<const int*>(base)[ScreenLayout::CNT + j] > 0) last = j;
        if (last >= 0) {
            const double* yp = reinterpret_cast<const double*>(base + ScreenLayout::SETS + last * ScreenLayout::SETSZ +
                                                               ScreenLayout::SET_Y);
            constexpr int Q = (SCREEN_CW_ROWS + 63) / 64;
            float y[Q], ymax = 0.f;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int row = lane + 64 * q;
                double t = 0.0;
                if (row < cw_rows)
                    for (int rp = 0; rp < SCREEN_YREPL; ++rp) t += yp[rp * SCREEN_CW_ROWS + row];
                y[q] = (float)t;
                ymax = fmaxf(ymax, y[q]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off));
            if (ymax > 0.f && ymax < 3.0e38f) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int row = lane + 64 * q;
                    if (row < cw_rows) base[ScreenLayout::CWV + row] = fmaxf(y[q] / ymax, 1e-3f);
                }
            }
        }
    }
    if (rec_host != nullptr) {
        __threadfence();          // this wave's own record writes are in memory before they are read back
        for (int i = threadIdx.x; i < 4 * SCREEN_MAX_SWEEPS + 4; i += 64)
            rec_host[i] = __hip_atomic_load(base + ScreenLayout::REC + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// After the last sweep of an enqueue: put the solve's result into the caller's buffer if it ended
// elsewhere.  The verdict records say where it ended: sweeps 0..done-1 count, the result is in
// buffer done % 3.  An inconclusive record (other than the last sweep's) means the host will finish the
// solve: nothing to do here.
__global__ __launch_bounds__(256) void screen_finalize_kernel(const float* base, int sub_iter, const float4* hb0,
                                                              const float4* hb1, const float4* hb2,
                                                              float4* __restrict__ dst, size_t n4) {
    __shared__ int src_s;
    if (threadIdx.x == 0) {
        int done = sub_iter, undecided = 0;
        for (int i = 0; i < sub_iter; ++i) {
            const int code = (int)base[ScreenLayout::REC + 4 * i + 2];
            if (code == 2) { done = i + 1; break; }
            // (the last sweep's test decides nothing - the loop ends after it either way - so an
            // inconclusive record there is as good as any)
            if (code != 1 && !(code == 3 && i + 1 == sub_iter)) { undecided = 1; break; }
        }
        src_s = undecided ? -1 : done % 3;
    }
    __syncthreads();
    if (src_s < 0) return;
    const float4* src = src_s == 0 ? hb0 : src_s == 1 ? hb1 : hb2;
    if (src == dst) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// Probe update after a power step: one workgroup of 256 threads; thread k owns row k of the
// r_pad <= 256 rows of Z.
__global__ __launch_bounds__(256) void screen_probe_update_kernel(float* base, int r) {
    __shared__ float Z[256 * 17];
    __shared__ float M[16 * 16];      // Z^T Z
    __shared__ float avec[16], mvec[16], nrm[16];
    const int tid = threadIdx.x;
    if (reinterpret_cast<const int*>(base)[ScreenLayout::FLAG] != 0) return;   // uniform

    // ---- gather the replicated accumulator (and clear it for the next power step) ----
    float zrow[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) zrow[p] = 0.f;
    for (int rep = 0; rep < SCREEN_ZREPL; ++rep) {
        float4* zp = reinterpret_cast<float4*>(base + ScreenLayout::ZACC + rep * 4096 + tid * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = zp[q];
            zrow[4 * q + 0] += v.x; zrow[4 * q + 1] += v.y; zrow[4 * q + 2] += v.z; zrow[4 * q + 3] += v.w;
            zp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (tid >= r) {
#pragma unroll
        for (int p = 0; p < 16; ++p) zrow[p] = 0.f;
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) Z[tid * 17 + p] = zrow[p];
    __syncthreads();

    // M = Z^T Z; a = its top eigenvector (a few power steps); t = Z a is the dominant direction of
    // the power-stepped subspace.  New probes: t, and Z_0..Z_14 with their component along t
    // removed; each column is then scaled by its measured norm, so every probe is a unit vector
    // (all the bound needs) whatever the rounding in between.
    {
        const int p = tid >> 4, q = tid & 15;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        for (int k = 0; k < 256; k += 4) {
            acc0 = fmaf(Z[k * 17 + p], Z[k * 17 + q], acc0);
            acc1 = fmaf(Z[(k + 1) * 17 + p], Z[(k + 1) * 17 + q], acc1);
            acc2 = fmaf(Z[(k + 2) * 17 + p], Z[(k + 2) * 17 + q], acc2);
            acc3 = fmaf(Z[(k + 3) * 17 + p], Z[(k + 3) * 17 + q], acc3);
        }
        M[p * 16 + q] = (acc0 + acc1) + (acc2 + acc3);
    }
    if (tid < 16) avec[tid] = base[ScreenLayout::AVEC + tid];
    __syncthreads();
    float trace = 0.f;
    for (int p = 0; p < 16; ++p) trace += M[p * 16 + p];
    if (!((trace > 0.f) && (trace < 3.0e38f))) return;        // no usable Z: keep the old probes (uniform)

    if (tid < 64) {                                           // lane p (mod 16) = component p
        float mrow[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) mrow[q] = M[(tid & 15) * 16 + q];
        float a = avec[tid & 15];
        float m = 0.f;
        for (int it = 0; it < 8; ++it) {
            m = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) m = fmaf(mrow[q], __shfl(a, q, 64), m);
            float n2 = m * m;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) n2 += __shfl_xor(n2, off);
            a = (n2 > 0.f) ? m * rsqrtf(n2) : a;
        }
        m = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) m = fmaf(mrow[q], __shfl(a, q, 64), m);
        float tt = a * m;                                     // t^T t = a^T M a
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) tt += __shfl_xor(tt, off);
        if (tid < 16) {
            avec[tid] = a;
            mvec[tid] = (tt > 0.f) ? m / tt : 0.f;            // c_q = (t^T Z_q) / (t^T t)
            base[ScreenLayout::AVEC + tid] = (tid == 0) ? 1.f : 0.f;   // next time: probe 0 is the top one
        }
    }
    __syncthreads();
    float urow[16];
    float trow = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) trow = fmaf(zrow[p], avec[p], trow);
    urow[0] = trow;
#pragma unroll
    for (int p = 1; p < 16; ++p) urow[p] = fmaf(-mvec[p - 1], trow, zrow[p - 1]);
    // measured column norms: squares into the (now free) Z tile, then 16 threads per column
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) Z[tid * 17 + p] = urow[p] * urow[p];
    __syncthreads();
    {
        const int p = tid >> 4, q = tid & 15;
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sacc += Z[(q + 16 * k) * 17 + p];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
        if (q == 0) nrm[p] = sacc;
    }
    __syncthreads();
    if (tid < 16 * ((r + 15) / 16)) {
        const int K = tid >> 4, jj = tid & 15;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float n2 = nrm[p];
            const float u = (n2 > 1e-30f && tid < r) ? urow[p] * rsqrtf(n2) * (1.0f - 1e-5f) : 0.f;
            base[ScreenLayout::UBLK + K * 256 + blk_off(p, jj)] = u;
        }
    }
}

// Collatz-Wielandt: for an entrywise nonnegative matrix M = H H^T and a positive vector v,
// lambda_max(M) <= max_k (M v)_k / v_k - a rigorous upper bound of sigma_max(H)^2 that is tight when v
// is close to the dominant eigenvector, which changes little from one solve to the next.
// y = H (H^T v): pass 1, thread = column: t_c = v . H[:, c] (coalesced); pass 2, thread = row: the
// workgroup's 256 columns of that row times t (contiguous per thread, served by the caches pass 1
// filled), one double atomic per row and workgroup.
__global__ __launch_bounds__(256) void screen_cw_kernel(int r, int b, const float* __restrict__ H, float* base) {
    __shared__ float vs[256];
    __shared__ float ts[256];
    const int tid = threadIdx.x;
    vs[tid] = tid < r ? base[ScreenLayout::CWV + tid] : 0.f;
    __syncthreads();
    const int c0 = blockIdx.x * 256;
    const int c = c0 + tid;
    float t = 0.f;
    if (c < b)
        for (int k = 0; k < r; ++k) t = fmaf(vs[k], H[(size_t)k * b + c], t);
    ts[tid] = t;
    __syncthreads();
    if (tid < r) {
        const float* row = H + (size_t)tid * b + c0;
        const int n = min(256, b - c0);
        float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
        int i = 0;
        if ((((uintptr_t)row) & 15) == 0) {
            for (; i + 3 < n; i += 4) {
                const float4 h = *reinterpret_cast<const float4*>(row + i);
                y0 = fmaf(h.x, ts[i], y0);
                y1 = fmaf(h.y, ts[i + 1], y1);
                y2 = fmaf(h.z, ts[i + 2], y2);
                y3 = fmaf(h.w, ts[i + 3], y3);
            }
        }
        for (; i < n; ++i) y0 = fmaf(row[i], ts[i], y0);
        atomicAdd(reinterpret_cast<double*>(base + ScreenLayout::CWY) + tid, (double)((y0 + y1) + (y2 + y3)));
    }
}

__global__ __launch_bounds__(256) void screen_cw_finish_kernel(int r, float* base, int k) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    double* yp = reinterpret_cast<double*>(base + ScreenLayout::CWY);
    const double y = tid < r ? yp[tid] : 0.0;
    const float v = tid < r ? base[ScreenLayout::CWV + tid] : 1.f;
    red[tid] = tid < r ? (float)(y / (double)v) * (1.0f + 1e-6f) : 0.f;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
        __syncthreads();
    }
    const float lam = red[0];
    __syncthreads();
    red[tid] = (float)y;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
        __syncthreads();
    }
    const float ymax = red[0];
    if (tid < r) {
        yp[tid] = 0.0;
        // the power step; kept strictly positive (rows of H that are all zero would zero it)
        if (ymax > 0.f && ymax < 3.0e38f) base[ScreenLayout::CWV + tid] = fmaxf((float)y / ymax, 1e-3f);
    }
    // (1 + 1e-4 on the square: fp32 rounding of the r- and 256-term sums behind y)
    if (tid == 0 && lam > 0.f && lam < 3.0e38f) base[ScreenLayout::UBH + k] = sqrtf(lam * (1.0f + 1e-4f)) * (1.0f + 1e-6f);
}


// ---------------------------------------------------------------------------------------------
// Device-side resolution of an inconclusive verdict (r <= 64).
//
// Where the bounds straddle stopping_diff the test has to be evaluated in full:
//     dist = sqrt(lambda_max(D D^T) / lambda_max(H H^T)),  D = H_new - H_old, H = H_old   (src/onmf.py:265)
// The host can do that (api.hip: exact_dist), at the price of two round trips through the launch queue
// and the PCIe bus per event; for small r - where a step is a few hundred microseconds and the small
// Gram matrices make borderline tests frequent - this kernel does it in place, enqueued ahead of time
// behind a one-launch solve and followed by another launch of the sweep kernel that takes its first
// sweep from device memory:
//   1. every workgroup finds the sweep i to settle in the verdict records (the first inconclusive one
//      with only "continue" before it; none: nothing to do), reads its columns of H_new (ring buffer
//      (i+1) % 3) and H_old (i % 3), stages D and H_old transposed in LDS and accumulates its part of the
//      two (four: below) RP x RP Gram matrices on the matrix cores (contraction over columns: the transposed tile puts
//      the column index on the MFMA's k lanes, and one fragment serves as both operands);
//   2. the parts go to a slab per workgroup; the workgroup that finishes last adds them in workgroup
//      order (bitwise reproducible) into LDS, in double;
//   3. two of its waves find the top eigenvalues in double by the host's own routine (Lanczos with full
//      re-orthogonalisation, the tridiagonal's top eigenvalue by multisection: lanczos_top_wave), one lane
//      makes the reference's comparison and writes the verdict; for "continue" it reopens the loop exactly
//      as the host would (screen_clear_flag_kernel): accumulator sets and counts of the next two sweeps
//      cleared, the exact sigma_max(H_old) + sigma_max(D) as the next sweep's bound, flag 0, NEXT = i + 1.
// A matrix with a NaN leaves the record as it is: the host then does it.
// For r <= 32 the kernel settles sweep i + 1 in the same go when sweep i's test says "go on": the launch that
// stopped had run sweep i + 1 ahead of the verdict, so its move is in the ring of three buffers too, and the
// two waves that would idle through step 3 take its two matrices; the loop then goes on at sweep i + 2.
// Measured (MI355X, d=441 r=25 b=1000, where every solve stops early and three of its tests - the stop and the
// two sweeps before it, 2 to 15 % above the threshold - need a full evaluation): the kernel takes 50 us, of
// which 30 or so are step 3 (tridiagonalisation 24, multisection 6: a lone wave issues an instruction every
// seven or eight clocks, and the algorithm is a chain of 23 dependent reflections), against about 60 us per
// test with the host in the loop.  Two rounds of (verdict kernel, this kernel, sweep launch) per solve: a step takes
// 0.23 ms against 0.30 with the host (0.315 with one sweep per round, i.e. three rounds).  At r = 64 (Lanczos,
// one sweep per round) a round costs about what the host's evaluation does.  Hence the default of
// ONMF_OPT_DEVICE_RESOLVE (api.hip).  (A power iteration in step 3 was far worse at small r: the moves' Gram
// matrix has a second eigenvalue of 0.7 of the first there and the iteration ran 200-600 steps.)
// ---------------------------------------------------------------------------------------------
constexpr int RESOLVE_COLS = 64;        // columns per staged tile
#ifdef ONMF_RESOLVE_MARKS            // diagnostic build: stage times of the resolving workgroup (100 MHz ticks)
#define RESOLVE_MARK(i) do { if (tid == 0) marks[i] = wall_clock64(); } while (0)
#else
#define RESOLVE_MARK(i) do {} while (0)
#endif

// wave-wide reductions on doubles: rows of 16 lanes by DPP rotations, the four row results by v_readlane
// (every lane ends with the same bits; an LDS-crossbar shuffle chain costs five times as much)
template <int CTRL>
__device__ __forceinline__ double dpp_rot_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int l) {          // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_add_f64(double v) {
    v += dpp_rot_f64<0x121>(v);
    v += dpp_rot_f64<0x122>(v);
    v += dpp_rot_f64<0x124>(v);
    v += dpp_rot_f64<0x128>(v);
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
// two sums at once: the two dependent chains fill each other's latencies
__device__ __forceinline__ void wave_add2_f64(double& a, double& b) {
    a += dpp_rot_f64<0x121>(a);
    b += dpp_rot_f64<0x121>(b);
    a += dpp_rot_f64<0x122>(a);
    b += dpp_rot_f64<0x122>(b);
    a += dpp_rot_f64<0x124>(a);
    b += dpp_rot_f64<0x124>(b);
    a += dpp_rot_f64<0x128>(a);
    b += dpp_rot_f64<0x128>(b);
    a = (lane_f64(a, 0) + lane_f64(a, 16)) + (lane_f64(a, 32) + lane_f64(a, 48));
    b = (lane_f64(b, 0) + lane_f64(b, 16)) + (lane_f64(b, 32) + lane_f64(b, 48));
}
// 1 / sqrt(x) and 1 / x for ordinary positive x (no denormals, no infinities): the hardware's estimate and two
// Newton steps - a third of the dependent instructions of the correctly rounded library routines
__device__ __forceinline__ double rsqrt_newton_f64(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}
__device__ __forceinline__ double rcp_newton_f64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_rot_f64<0x121>(v));
    v = fmax(v, dpp_rot_f64<0x122>(v));
    v = fmax(v, dpp_rot_f64<0x124>(v));
    v = fmax(v, dpp_rot_f64<0x128>(v));
    return fmax(fmax(lane_f64(v, 0), lane_f64(v, 16)), fmax(lane_f64(v, 32), lane_f64(v, 48)));
}
// what one phase wrote to LDS is read by other lanes of the same wave in the next
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Top eigenvalue of the symmetric tridiagonal of order m <= 64 by multisection, one wave.  Lane i holds the
// diagonal entry a_i and the entry b_i that couples rows i and i + 1 (what lanes >= m and b_{m-1} hold is not
// used).  Every round the 64 lanes count the eigenvalues below 64 points that divide the bracket evenly - sign
// changes in the sequence of characteristic polynomials of the leading blocks,
//     p_i = (a_i - x) p_{i-1} - b_{i-1}^2 p_{i-2},
// one multiply-add on the dependent chain per step, the coefficient pair of the step broadcast from LDS (coef:
// 65 pairs of doubles, 16-byte aligned), the sign bits shifted into a word and counted every eighth step, where
// the values are also rescaled so that they neither overflow nor vanish (the matrix is scaled to norm <= 1
// first).  The recurrence runs in whole blocks of eight steps: the tridiagonal is padded with decoupled rows
// far below the bracket, each of which adds one to every lane's count.  The bracket shrinks to the
// sub-interval in which the count reaches the order - 65 times narrower per round, down to a relative width of
// TRIDIAG_REL_WIDTH after eight.  `floor`: a known lower bound, or -inf.
constexpr double TRIDIAG_REL_WIDTH = 1e-14;
constexpr int TRIDIAG_COEF_DOUBLES = 2 * 65;
__device__ double tridiag_top_lanes(double a_mine, double b_mine, int m, double floor, int lane, double* coef) {
    if (m == 1) return lane_f64(a_mine, 0);
    const double b_prev = __shfl_up(b_mine, 1);                    // lane i gets lane i - 1's
    const double off = (lane > 0 && lane < m ? fabs(b_prev) : 0.0) + (lane + 1 < m ? fabs(b_mine) : 0.0);
    const double a_in = lane < m ? a_mine : lane_f64(a_mine, 0);
    const double off_in = lane < m ? off : 0.0;
    double lo = -wave_max_f64(off_in - a_in);                       // Gershgorin
    double hi = wave_max_f64(a_in + off_in);
    // the top eigenvalue is at least the largest diagonal entry
    lo = fmax(lo, wave_max_f64(a_in));
    if (floor > lo && floor <= hi) lo = floor;
    const double norm = fmax(fabs(lo), fabs(hi));
    if (!(norm > 0.0)) return 0.0;
    const double inv = 1.0 / norm;
    lo *= inv;
    hi *= inv;
    const int steps = (m - 1 + 7) / 8 * 8;                          // i = 1 .. steps, rows m .. steps are padding
    double2* cp = reinterpret_cast<double2*>(coef);
    cp[lane] = (lane >= 1 && lane < m) ? make_double2(a_mine * inv, (b_prev * inv) * (b_prev * inv))
                                       : make_double2(-10.0, 0.0);
    if (lane == 0) cp[64] = make_double2(-10.0, 0.0);
    const double a0 = lane_f64(a_mine, 0) * inv;
    wave_lds_fence();
    for (int round = 0; round < 16; ++round) {
        const double width = hi - lo;
        if (!(width > TRIDIAG_REL_WIDTH)) break;
        const double x = lo + width * ((double)(lane + 1) * (1.0 / 65.0));
        double p0 = 1.0, p1 = a0 - x;
        unsigned w = (unsigned)__double2hiint(p1) >> 31;            // bit 0: the last value's sign
        int cnt = (int)w;                                           // (p_0 = 1 is positive)
        bool sawzero = p1 == 0.0;
        for (int i0 = 1; i0 <= steps; i0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double2 c = cp[i0 + j];
                const double p2 = fma(c.x - x, p1, -c.y * p0);
                p0 = p1;
                p1 = p2;
                w = __builtin_amdgcn_alignbit(w, (unsigned)__double2hiint(p1), 31);      // (w << 1) | sign
            }
            cnt += __popc((w ^ (w >> 1)) & 0xFFu);
            // (a value of exactly zero counts as either sign - its successor has the sign opposite to its
            // predecessor's, so the count is the same - unless the coupling entry is zero too: then it stays
            // zero to the end of the block and the careful form below takes over)
            sawzero |= p1 == 0.0;
            const double mag = fabs(p1);
            const double sc = mag > 1e100 ? 1e-100 : (mag < 1e-100 ? 1e100 : 1.0);
            p0 *= sc;
            p1 *= sc;
        }
        int order = steps + 1;
        if (__ballot(sawzero) != 0ull) {
            // count again, zeros given the sign of their predecessor, without the padding
            p0 = 1.0;
            p1 = a0 - x;
            bool neg = p1 < 0.0;
            cnt = neg ? 1 : 0;
            for (int i = 1; i < m; ++i) {
                const double2 c = cp[i];
                const double p2 = fma(c.x - x, p1, -c.y * p0);
                p0 = p1;
                p1 = p2;
                const double mag = fabs(p1);
                const double sc = mag > 1e100 ? 1e-100 : (mag < 1e-100 ? 1e100 : 1.0);
                p0 *= sc;
                p1 *= sc;
                if (p1 != 0.0) {
                    const bool n2 = p1 < 0.0;
                    cnt += (n2 != neg) ? 1 : 0;
                    neg = n2;
                }
            }
            order = m;
        }
        // lanes below the top eigenvalue count fewer than the order: they form a prefix
        const unsigned long long below = __ballot(cnt < order);
        const int k = below == ~0ull ? 64 : __ffsll((long long)~below) - 1;
        const double nlo = k > 0 ? lane_f64(x, k - 1) : lo;
        const double nhi = k < 64 ? lane_f64(x, k) : hi;
        if (nlo == lo && nhi == hi) break;              // down to neighbouring doubles
        lo = nlo;
        hi = nhi;
    }
    wave_lds_fence();
    return 0.5 * (lo + hi) * norm;
}
// the same with the coefficients in LDS arrays
__device__ double tridiag_top_wave(const double* a, const double* b, int m, double floor, int lane, double* coef) {
    return tridiag_top_lanes(lane < m ? a[lane] : 0.0, lane + 1 < m ? b[lane] : 0.0, m, floor, lane, coef);
}

// Largest eigenvalue of the symmetric positive semi-definite n x n matrix M (doubles in LDS, rows RP apart),
// one wave: Lanczos with full re-orthogonalisation (classical Gram-Schmidt, twice) and the tridiagonal's top
// eigenvalue by multisection - the routine the host runs on the same matrices (host_eig.cpp), lane i
// owning component i.  All n steps for n <= 32 (exact, whatever the spectrum: at r = 16-25 the moves'
// Gram matrix has a second eigenvalue of 0.7 of the first, which a power iteration needs hundreds of steps
// for); above, the run ends when the top Ritz value has stopped moving (looked at every fourth step).
// work: the multisection's coefficient pairs, (n + 1) x LZ_PITCH doubles of Lanczos vectors, 64 + 64 + 64 + 64 doubles.
constexpr int LZ_PITCH = 65;       // rows of the vector store one double more than a wave apart: lane l reading
                                   // its own row meets no other lane's bank
__device__ double lanczos_top_wave(const double* M, int n, int RP, double* work, int lane, int* ok) {
    double* coef = work;               // (first: 16-byte aligned)
    double* V = work + TRIDIAG_COEF_DOUBLES;
    double* wb = V + (size_t)(n + 1) * LZ_PITCH;
    double* cb = wb + 64;
    double* al = cb + 64;
    double* be = al + 64;
    const bool act = lane < n;
    *ok = 1;
    double f = 0.0;
    if (act)
        for (int j = 0; j < n; ++j) f = fma(M[j * RP + lane], M[j * RP + lane], f);
    const double fro = wave_add_f64(f);
    if (!(fro > 0.0)) {
        *ok = (fro == 0.0);
        return 0.0;
    }
    if (n == 1) return M[0];
    double v = act ? 1.0 + 0.5 * sin(12.9898 * (double)(lane + 1)) : 0.0;
    v *= 1.0 / sqrt(wave_add_f64(v * v));
    V[lane] = v;
    wave_lds_fence();
    const double ninf = -__builtin_inf();
    double prev = -1.0;
    int m = 0;
    for (int j = 0; j < n; ++j) {
        const double* vj = V + j * LZ_PITCH;
        double w = 0.0;
        if (act) {
#pragma unroll 8
            for (int i = 0; i < n; ++i) w = fma(M[i * RP + lane], vj[i], w);     // M symmetric: column = row
        }
        const double alpha = wave_add_f64(w * v);
        for (int pass = 0; pass < 2; ++pass) {
            wb[lane] = w;
            wave_lds_fence();
            double cl = 0.0;
            if (lane <= j) {
#pragma unroll 8
                for (int i = 0; i < n; ++i) cl = fma(V[lane * LZ_PITCH + i], wb[i], cl);
            }
            cb[lane] = cl;
            wave_lds_fence();
            if (act) {
#pragma unroll 8
                for (int l = 0; l <= j; ++l) w = fma(-cb[l], V[l * LZ_PITCH + lane], w);
            }
            wave_lds_fence();
        }
        const double bn = sqrt(wave_add_f64(w * w));
        if (lane == 0) al[j] = alpha;
        m = j + 1;
        wave_lds_fence();
        if (bn <= 1e-14 * sqrt(fro) || m == n) break;   // invariant subspace, or all n steps: exact
        if (n > 32 && m >= 8 && (m & 3) == 0) {
            const double theta = tridiag_top_wave(al, be, m, prev > 0.0 ? prev : ninf, lane, coef);
            if (prev > 0.0 && theta - prev <= 1e-13 * theta) return theta;
            prev = theta;
        }
        if (lane == 0) be[j] = bn;
        v = w * (1.0 / bn);
        V[(j + 1) * LZ_PITCH + lane] = v;
        wave_lds_fence();
    }
    return tridiag_top_wave(al, be, m, prev > 0.0 ? prev : ninf, lane, coef);
}

// The same value for small matrices (n <= N, N = 16 or 32) by Householder tridiagonalisation, one wave: n - 2
// reflections, each a product with the trailing matrix, two wave-wide sums and a symmetric rank-two update -
// exact after a fixed number of operations whatever the spectrum, and a fraction of what the Lanczos run with
// its re-orthogonalisation costs at these sizes.  Lane j keeps column j of the matrix in registers (rows and
// columns from n on are zero); entries of the reflector and of the update's vector travel by v_readlane, so
// the loops touch no memory - at the price of straight-line code, unrolled over the reflections, that a lone
// wave fetches about as slowly as it runs it (24 us at n = 25 against 32 us for the same steps with the matrix
// in LDS and rolled loops, whose every batch of rows waits for an LDS round trip).
// M: n x n doubles in LDS, rows RP apart (read once).  work: the multisection's coefficient pairs.
template <int N>
__device__ double householder_top_wave(const double* M, int n, int RP, double* work, int lane, int* ok) {
    *ok = 1;
    double m[N];
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = (lane < n && i < n) ? M[i * RP + lane] : 0.0;
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) f = fma(m[i], m[i], f);
    const double fro = wave_add_f64(f);
    if (!(fro > 0.0)) {
        *ok = (fro == 0.0);
        return 0.0;
    }
    if (n == 1) return lane_f64(m[0], 0);
    double a_mine = 0.0, b_mine = 0.0;                              // lane k: a_k and b_k of the tridiagonal
#pragma unroll
    for (int k = 0; k + 1 < N; ++k) {
        if (k + 1 >= n) continue;                                   // (wave-uniform)
        if (lane == k) a_mine = m[k];
        const double x = lane > k ? m[k] : 0.0;                     // row k = column k (symmetric)
        const double x0 = lane_f64(x, k + 1);
        double bk = x0;
        if (k + 2 < n) {
            // y = S x over the trailing rows, before the reflector's scale is known: S u = S x - alpha S e_{k+1}
            // and u^T S u = x^T S x - 2 alpha (S x)_{k+1} + alpha^2 S_{k+1,k+1}, so both sums of the step (|x|^2
            // and x^T S x) travel through the wave together, and the square root waits for one reduction only
            double y4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = k + 1; i < N; ++i)
                if ((i & ~3) < n) y4[i & 3] = fma(m[i], lane_f64(x, i), y4[i & 3]);      // (rows from n on are zero)
            const double y = (y4[0] + y4[1]) + (y4[2] + y4[3]);
            double nx2 = x * x, xy = x * y;
            wave_add2_f64(nx2, xy);
            if (nx2 > x0 * x0) {                                    // something below the sub-diagonal: reflect
                const double nx = nx2 * rsqrt_newton_f64(nx2);
                const double alpha = x0 > 0.0 ? -nx : nx;
                const double tau = rcp_newton_f64(fma(nx, fabs(x0), nx2));     // 2 / |u|^2,  u = x - alpha e_{k+1}
                const double usu = fma(alpha * alpha, lane_f64(m[k + 1], k + 1), fma(-2.0 * alpha, lane_f64(y, k + 1), xy));
                const double u = (lane == k + 1) ? x - alpha : x;
                const double pj = tau * fma(-alpha, m[k + 1], y);   // tau S u  (lane j: S_{j,k+1} = S_{k+1,j} = m[k+1])
                const double q = fma(-0.5 * tau * tau * usu, u, pj);
#pragma unroll
                for (int i = k + 1; i < N; ++i)
                    if ((i & ~3) < n) m[i] = fma(-lane_f64(u, i), q, fma(-lane_f64(q, i), u, m[i]));
                bk = alpha;
            }
        }
        if (lane == k) b_mine = bk;
        if (k + 2 == n && lane == k + 1) a_mine = m[k + 1];
    }
    return tridiag_top_lanes(a_mine, b_mine, n, -__builtin_inf(), lane, work);
}

// work space of one wave's eigenvalue run, in doubles (even: its (u, q) pairs are read 16 bytes at a time)
constexpr int eig_work_doubles(int RP) { return (TRIDIAG_COEF_DOUBLES + (RP + 1) * LZ_PITCH + 4 * 64 + 1) & ~1; }

template <int NB>
__global__ __launch_bounds__(256) void screen_resolve_kernel(float* base, int r, int b, const float* hb0,
                                                             const float* hb1, const float* hb2, int sub_iter,
                                                             float tol, float* slab, int cols_per_wg) {
    constexpr int RP = 16 * NB;
    constexpr int PITCH = RP + 4;                       // staged tiles are [column][row], rows padded
    constexpr int NPAIR = NB * NB;                      // 16 x 16 output blocks of one Gram matrix
    constexpr int PER_WAVE = (NPAIR + 3) / 4;
    constexpr int EIG_WORK = eig_work_doubles(RP);
    // r <= 32: the sweep after the inconclusive one is settled in the same go (its move is in the ring too: the
    // stopped launch ran it ahead of the verdict), by the two waves that would otherwise idle through step 3
    constexpr bool TWO = NB <= 2;
    constexpr int NM = TWO ? 4 : 2;                     // matrices: D_i D_i^T, H_i H_i^T (, D_{i+1} .., H_{i+1} ..)
    constexpr int TILE = RESOLVE_COLS * PITCH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int sel_s, last_s;
    __shared__ double lam_s[4];
    __shared__ int ok_s[4];
    int* ibase = reinterpret_cast<int*>(base);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
#ifdef ONMF_RESOLVE_MARKS
    long long marks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RESOLVE_MARK(0);
    if (tid == 0) {
        int sel = -1;
        if (ibase[ScreenLayout::FLAG] != 0) {           // the loop was ended: by a stop, or for want of a verdict
            for (int i = 0; i < sub_iter; ++i) {
                const int code = (int)base[ScreenLayout::REC + 4 * i + 2];
                if (code == 1) continue;
                if (code == 3 && i + 1 < sub_iter) sel = i;   // (the last sweep's test decides nothing)
                break;
            }
        }
        sel_s = sel;
    }
    __syncthreads();
    const int isel = sel_s;
    if (isel < 0) {
        if (blockIdx.x == 0 && tid == 0) ibase[ScreenLayout::NEXT] = sub_iter;    // nothing follows
        return;
    }
    const bool two = TWO && isel + 2 < sub_iter;        // (sweep isel + 1 ran, and its test decides something)
    const int nm = two ? 4 : 2;
    auto ring = [&](int k) { return (k % 3 == 0) ? hb0 : (k % 3 == 1) ? hb1 : hb2; };
    const float* H0p = ring(isel);                      // code before sweep isel, after it, after the next
    const float* H1p = ring(isel + 1);
    const float* H2p = ring(isel + 2);

    RESOLVE_MARK(1);
    // ---- 1. this workgroup's part of the Gram matrices ----
    f32x4 acc[NM][PER_WAVE];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int p = 0; p < PER_WAVE; ++p) acc[m][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col_lo = blockIdx.x * cols_per_wg, col_hi = min(b, col_lo + cols_per_wg);
    // (a tile's loads are all issued before the first is used, and the next tile's before this one's products:
    // one round trip to memory per tile, not one per element)
    constexpr int PER = RP * RESOLVE_COLS / 256;        // elements of a tile per thread and matrix
    float h0[PER], h1[PER], h2[TWO ? PER : 1];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q;
            const int row = e / RESOLVE_COLS, col = c0 + e % RESOLVE_COLS;   // consecutive threads: consecutive columns
            const bool in = row < r && col < col_hi;
            h0[q] = in ? H0p[(size_t)row * b + col] : 0.f;
            h1[q] = in ? H1p[(size_t)row * b + col] : 0.f;
            if constexpr (TWO) h2[q] = (in && two) ? H2p[(size_t)row * b + col] : 0.f;
        }
    };
    if (col_lo < col_hi) fetch(col_lo);
    for (int c0 = col_lo; c0 < col_hi; c0 += RESOLVE_COLS) {
        __syncthreads();                                // the previous tile's readers are done
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + 256 * q;
            const int row = e / RESOLVE_COLS, cc = e % RESOLVE_COLS;
            lds[0 * TILE + cc * PITCH + row] = h1[q] - h0[q];
            lds[1 * TILE + cc * PITCH + row] = h0[q];
            if constexpr (TWO) {
                lds[2 * TILE + cc * PITCH + row] = h2[q] - h1[q];
                lds[3 * TILE + cc * PITCH + row] = h1[q];
            }
        }
        if (c0 + RESOLVE_COLS < col_hi) fetch(c0 + RESOLVE_COLS);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m >= nm) continue;                      // (uniform)
            const float* T = lds + m * TILE;
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) {
                const int pair = wave + 4 * p;
                if (pair < NPAIR) {                     // wave-uniform
                    const int bi = pair / NB, bj = pair % NB;
#pragma unroll
                    for (int q = 0; q < RESOLVE_COLS / 4; ++q) {
                        // fragment of block row bi: lane (m = c16, k = g4) = X[16 bi + m][column 4 q + k]; the
                        // same layout serves as the B operand of X X^T with n = c16
                        const int at = (4 * q + g4) * PITCH + c16;
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[at + 16 * bi], T[at + 16 * bj], acc[m][p], 0, 0, 0);
                    }
                }
            }
        }
    }
    RESOLVE_MARK(2);
    // ---- 2. slab of this workgroup, then the last one adds them all ----
    float* mine = slab + (size_t)blockIdx.x * NM * RP * RP;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if (m >= nm) continue;
#pragma unroll
        for (int p = 0; p < PER_WAVE; ++p) {
            const int pair = wave + 4 * p;
            if (pair < NPAIR) {
                const int bi = pair / NB, bj = pair % NB;
#pragma unroll
                for (int t = 0; t < 4; ++t) {           // acc[t] = S[16 bi + 4 g + t][16 bj + c]
                    const int at = (16 * bi + 4 * g4 + t) * RP + 16 * bj + c16;
                    __hip_atomic_store(mine + m * RP * RP + at, acc[m][p][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const int done = __hip_atomic_fetch_add(ibase + ScreenLayout::RCNT, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_s = (done == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    RESOLVE_MARK(3);
    // (the staging tiles are dead: the double-precision matrices take their place)
    double* S = reinterpret_cast<double*>(lds);         // [NM][RP * RP]
    // (the other workgroups released their slabs before they counted themselves in, and the fence above
    // is the acquire: ordinary vector loads from here on, many in flight)
    {
        constexpr int QUADS = NM * RP * RP / 4;
        const int quads = nm * RP * RP / 4;
        const float4* slab4 = reinterpret_cast<const float4*>(slab);
        for (int e = tid; e < quads; e += 256) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            unsigned w = 0;
            for (; w + 7 < gridDim.x; w += 8) {             // (eight slabs' loads in flight: one round trip)
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = slab4[(size_t)(w + j) * QUADS + e];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s0 += (double)v[j].x; s1 += (double)v[j].y; s2 += (double)v[j].z; s3 += (double)v[j].w;
                }
            }
            for (; w + 3 < gridDim.x; w += 4) {
                const float4 a = slab4[(size_t)(w + 0) * QUADS + e], bq = slab4[(size_t)(w + 1) * QUADS + e];
                const float4 cq = slab4[(size_t)(w + 2) * QUADS + e], dq = slab4[(size_t)(w + 3) * QUADS + e];
                s0 += (double)a.x; s1 += (double)a.y; s2 += (double)a.z; s3 += (double)a.w;
                s0 += (double)bq.x; s1 += (double)bq.y; s2 += (double)bq.z; s3 += (double)bq.w;
                s0 += (double)cq.x; s1 += (double)cq.y; s2 += (double)cq.z; s3 += (double)cq.w;
                s0 += (double)dq.x; s1 += (double)dq.y; s2 += (double)dq.z; s3 += (double)dq.w;
            }
            for (; w < gridDim.x; ++w) {
                const float4 a = slab4[(size_t)w * QUADS + e];
                s0 += (double)a.x; s1 += (double)a.y; s2 += (double)a.z; s3 += (double)a.w;
            }
            S[4 * e] = s0; S[4 * e + 1] = s1; S[4 * e + 2] = s2; S[4 * e + 3] = s3;
        }
    }
    __syncthreads();
    RESOLVE_MARK(4);
    // ---- 3. top eigenvalues: one wave per matrix ----
    if (wave < nm) {
        double* work = S + NM * RP * RP + (size_t)wave * EIG_WORK;
        int ok = 0;
        double* M = S + wave * RP * RP;
        double theta;
        if constexpr (RP <= 16) theta = householder_top_wave<16>(M, r, RP, work, lane, &ok);
        else if constexpr (RP <= 32) theta = householder_top_wave<32>(M, r, RP, work, lane, &ok);
        else theta = lanczos_top_wave(M, r, RP, work, lane, &ok);
        if (lane == 0) {
            lam_s[wave] = theta;
            ok_s[wave] = ok;
        }
    }
    __syncthreads();
    RESOLVE_MARK(5);
    const double lD = lam_s[0], lH = lam_s[1];
    const bool settled = ok_s[0] && ok_s[1] && lH > 0.0 && lD >= 0.0;     // else: left to the host
    const double dist = settled ? sqrt(lD) / sqrt(lH) : 0.0;
    const bool keep = dist > (double)tol;               // the loop test of src/onmf.py:252
    double lD1 = 0.0, lH1 = 0.0, dist1 = 0.0;
    bool settled1 = false, keep1 = false;
    if (two && settled && keep) {
        lD1 = lam_s[2];
        lH1 = lam_s[3];
        settled1 = ok_s[2] && ok_s[3] && lH1 > 0.0 && lD1 >= 0.0;
        dist1 = settled1 ? sqrt(lD1) / sqrt(lH1) : 0.0;
        keep1 = dist1 > (double)tol;
    }
    // where the loop goes on: at sweep isel + 1 when only sweep isel was settled (the stopped launch had run
    // that sweep once ahead of the verdict: it runs again), at isel + 2 when the sweep after it was too
    // (its output is in the ring); nowhere after a stop
    const int reopen = (settled && keep) ? (settled1 ? (keep1 ? isel + 2 : -1) : isel + 1) : -1;
    if (reopen >= 0) {
        for (int j = reopen; j < SCREEN_MAX_SWEEPS && j < reopen + 2; ++j) {
            for (int e = tid; e < ScreenLayout::SETSZ; e += 256) base[ScreenLayout::SETS + j * ScreenLayout::SETSZ + e] = 0.f;
            if (tid == 0) ibase[ScreenLayout::CNT + j] = 0;
        }
    }
    __syncthreads();
#ifdef ONMF_RESOLVE_MARKS
    if (tid == 0)
        printf("resolve marks (10 ns): find %lld, gram %lld, slab+count %lld, add slabs %lld, eig %lld\n", marks[1] - marks[0],
               marks[2] - marks[1], marks[3] - marks[2], marks[4] - marks[3], marks[5] - marks[4]);
#endif
    if (tid == 0) {
        ibase[ScreenLayout::RCNT] = 0;
        if (settled) {
            float* rec = base + ScreenLayout::REC + 4 * isel;
            rec[0] = (float)dist;
            rec[1] = (float)dist;
            rec[2] = keep ? 1.0f : 2.0f;
            if (settled1) {
                rec[4] = (float)dist1;
                rec[5] = (float)dist1;
                rec[6] = keep1 ? 1.0f : 2.0f;
            }
            ibase[ScreenLayout::NRES] += settled1 ? 2 : 1;
            ibase[ScreenLayout::RNDS] += 1;
            if (reopen >= 0) {
                const double ub = reopen == isel + 2 ? sqrt(lH1) + sqrt(lD1) : sqrt(lH) + sqrt(lD);
                base[ScreenLayout::UBH + reopen] = (float)(ub * (1.0 + 1e-5));
                ibase[ScreenLayout::NEXT] = reopen;
                __threadfence();
                ibase[ScreenLayout::FLAG] = 0;
            } else {
                ibase[ScreenLayout::NEXT] = sub_iter;
            }
        } else {
            ibase[ScreenLayout::NEXT] = sub_iter;
        }
    }
}

template <int NB>
int run_resolve(onmf_ctx* c, float* base, int r, int b, float* const* hb, int sub_iter, float tol, hipStream_t s) {
    constexpr int RP = 16 * NB;
    constexpr int NM = NB <= 2 ? 4 : 2;
    // columns per workgroup: a multiple of the tile, at most 32 workgroups (the one that finishes last adds
    // up all their slabs)
    int cols = std::max(RESOLVE_COLS, (b + 31) / 32);
    cols = (cols + RESOLVE_COLS - 1) / RESOLVE_COLS * RESOLVE_COLS;
    const int nwg = (b + cols - 1) / cols;
    ONMF_TRY(onmf_reserve(c, c->rslab, (size_t)nwg * NM * RP * RP * sizeof(float), "verdict resolution slabs"));
    const size_t tiles = (size_t)NM * RESOLVE_COLS * (RP + 4) * sizeof(float);
    const size_t mats = ((size_t)NM * RP * RP + NM * eig_work_doubles(RP)) * sizeof(double);
    const size_t lds = std::max(tiles, mats);
    static LdsLimit lim;
    ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)screen_resolve_kernel<NB>, lds));
    hipLaunchKernelGGL((screen_resolve_kernel<NB>), dim3(nwg), dim3(256), lds, s, base, r, b, hb[0], hb[1], hb[2],
                       sub_iter, tol, (float*)c->rslab.p, cols);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

}  // namespace

int launch_screen_cw(onmf_ctx* c, float* base, int r, int b, const float* H, int k, hipStream_t s) {
    if (r > 256) return onmf_fail(c, ONMF_ERR_ARG, "screen bound: r=%d > 256", r);
    hipLaunchKernelGGL(screen_cw_kernel, dim3((b + 255) / 256), dim3(256), 0, s, r, b, H, base);
    hipLaunchKernelGGL(screen_cw_finish_kernel, dim3(1), dim3(256), 0, s, r, base, k);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_screen_init(onmf_ctx* c, float* base, int r, hipStream_t s, int n_sets, float* fill_out, size_t fill_n,
                       uint64_t fill_seed) {
    const int keep = (c->screen_r == r) ? 1 : 0;
    c->screen_r = r;
    n_sets = std::max(1, std::min(n_sets, SCREEN_MAX_SWEEPS));
    unsigned fill_blocks = 0;
    if (fill_out != nullptr && fill_n > 0)
        fill_blocks = (unsigned)std::min<size_t>(((fill_n + 3) / 4 + 255) / 256, 4096);
    hipLaunchKernelGGL(screen_init_kernel, dim3(1 + fill_blocks), dim3(256), 0, s, base, r, keep, n_sets, fill_out,
                       fill_n, fill_seed);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_screen_final(onmf_ctx* c, float* base, int k_from, int k_last, float tol, float rank_cap,
                        hipStream_t s, const int* start_dev, int cw_rows, float* rec_host) {
    hipLaunchKernelGGL(screen_final_kernel, dim3(1), dim3(64), 0, s, base, k_from, k_last, tol, rank_cap, start_dev,
                       cw_rows, rec_host);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_screen_finalize(onmf_ctx* c, const float* base, int sub_iter, float* const* hb, float* dst,
                           size_t n, hipStream_t s) {
    if ((n & 3) || (((uintptr_t)hb[0] | (uintptr_t)hb[1] | (uintptr_t)hb[2] | (uintptr_t)dst) & 15))
        return onmf_fail(c, ONMF_ERR_ARG, "screen finalize: buffers must be 16-byte aligned multiples of 4 floats");
    hipLaunchKernelGGL(screen_finalize_kernel, dim3(1024), dim3(256), 0, s, base, sub_iter, (const float4*)hb[0],
                       (const float4*)hb[1], (const float4*)hb[2], (float4*)dst, n / 4);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_screen_refresh(onmf_ctx* c, float* base, int r, int b, const float* Hnew, const float* Hold,
                          hipStream_t s) {
    int rc = ONMF_ERR_ARG;
    switch ((r + 15) / 16) {
#define RCASE(N) case N: rc = run_refresh<N>(c, r, b, Hnew, Hold, base, s); break
        RCASE(1); RCASE(2); RCASE(3); RCASE(4); RCASE(5); RCASE(6); RCASE(7); RCASE(8);
        RCASE(9); RCASE(10); RCASE(11); RCASE(12); RCASE(13); RCASE(14); RCASE(15); RCASE(16);
#undef RCASE
        default: return onmf_fail(c, ONMF_ERR_ARG, "screen refresh: unsupported r=%d", r);
    }
    if (rc != ONMF_OK) return rc;
    hipLaunchKernelGGL(screen_probe_update_kernel, dim3(1), dim3(256), 0, s, base, r);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

bool screen_resolve_covers(int r) { return r >= 1 && r <= 64; }

int launch_screen_resolve(onmf_ctx* c, float* base, int r, int b, float* const* hb, int sub_iter, float tol,
                          hipStream_t s) {
    switch ((r + 15) / 16) {
        case 1: return run_resolve<1>(c, base, r, b, hb, sub_iter, tol, s);
        case 2: return run_resolve<2>(c, base, r, b, hb, sub_iter, tol, s);
        case 3: return run_resolve<3>(c, base, r, b, hb, sub_iter, tol, s);
        case 4: return run_resolve<4>(c, base, r, b, hb, sub_iter, tol, s);
    }
    return onmf_fail(c, ONMF_ERR_ARG, "verdict resolution on the device covers r <= 64 (r=%d)", r);
}

int launch_screen_clear_flag(onmf_ctx* c, float* base, int k, hipStream_t s, float ub) {
    hipLaunchKernelGGL(screen_clear_flag_kernel, dim3(1), dim3(64), 0, s, base, k, ub);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}
