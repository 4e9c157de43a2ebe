// Small bandwidth-bound kernels around the ONMF step on gfx950: uniform fill (the device stand-in
// for np.random.rand, src/onmf.py:190/:246), column gather X[:, idx] (onmf.py:214), the
// code[:, idx] += H accumulation with NumPy's duplicate-index semantics (onmf.py:221), the mode
// unfolding of a 3-way tensor (src/ontf.py:204-207) and a tiled transpose.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void fill_uniform_kernel(float* __restrict__ out, size_t n,
                                                            uint64_t seed) {
    const size_t nquads = (n + 3) / 4;
    for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < nquads;
         q += (size_t)gridDim.x * blockDim.x)
        fill_uniform_quad(out, n, q, seed);
}

__global__ __launch_bounds__(256) void gather_columns_kernel(int d, int n, const float* __restrict__ X,
                                                             int ldx, const long long* __restrict__ idx,
                                                             int b, float* __restrict__ Xb) {
    const size_t total = (size_t)d * b;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / b), col = (int)(i % b);
        long long src = idx[col];
        if (src < 0) src += n;                 // NumPy negative indexing
        Xb[i] = (src >= 0 && src < n) ? X[(size_t)row * ldx + src] : 0.f;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                   float a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        y[i] = fmaf(a, x[i], y[i]);
}

// out = keep * old + w * part   (the aggregate recursion applied to sums formed elsewhere)
__global__ __launch_bounds__(256) void axpby_kernel(float* out, const float* old, float keep,
                                                    const float* __restrict__ part, float w, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = fmaf(keep, old[i], w * part[i]);
}

__global__ __launch_bounds__(256) void sub_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                  const float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] - b[i];
}

// out[0 .. n) = a - b and out[n .. 2n) = b: the two operands of the stopping test stacked, so that ONE
// product of the stack with its own transpose holds both Gram matrices
__global__ __launch_bounds__(256) void sub_stack_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                        const float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float bv = b[i];
        out[i] = a[i] - bv;
        out[n + i] = bv;
    }
}

// last[j] = largest position p with idx[p] == j (or -1)
__global__ __launch_bounds__(256) void last_pos_init_kernel(int* last, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) last[i] = -1;
}
__global__ __launch_bounds__(256) void last_pos_kernel(int* last, int n, const long long* __restrict__ idx,
                                                       int b) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < b; p += gridDim.x * blockDim.x) {
        long long j = idx[p];
        if (j < 0) j += n;
        if (j >= 0 && j < n) atomicMax(&last[j], p);
    }
}
__global__ __launch_bounds__(256) void scatter_last_kernel(int r, int n, float* __restrict__ code,
                                                           const int* __restrict__ last,
                                                           const long long* __restrict__ idx, int b,
                                                           const float* __restrict__ H) {
    const size_t total = (size_t)r * b;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / b), p = (int)(i % b);
        long long j = idx[p];
        if (j < 0) j += n;
        if (j >= 0 && j < n && last[j] == p) code[(size_t)row * n + j] += H[i];
    }
}

// out[a][rest] with a = index along `mode`, rest = remaining two axes in C order; or transposed.
__global__ __launch_bounds__(256) void unfold3_kernel(int n0, int n1, int n2, const float* __restrict__ T,
                                                      int mode, int transpose, float* __restrict__ out) {
    const size_t total = (size_t)n0 * n1 * n2;
    const int dims[3] = {n0, n1, n2};
    const int nm = dims[mode];
    const size_t rest = total / nm;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total;
         o += (size_t)gridDim.x * blockDim.x) {
        // o indexes the OUTPUT so that stores are coalesced
        size_t a, q;
        if (!transpose) { a = o / rest; q = o % rest; } else { q = o / nm; a = o % nm; }
        size_t i0, i1, i2;
        if (mode == 0) { i0 = a; i1 = q / n2; i2 = q % n2; }
        else if (mode == 1) { i1 = a; i0 = q / n2; i2 = q % n2; }
        else { i2 = a; i0 = q / n1; i1 = q % n1; }
        out[o] = T[(i0 * n1 + i1) * n2 + i2];
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const float* __restrict__ in,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    __builtin_amdgcn_s_setprio(3);     // small and often in a latency chain beside a chip-filling kernel
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int rr = by + k, cc = bx + tx;
        tile[k][tx] = (rr < rows && cc < cols) ? in[(size_t)rr * cols + cc] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int cc = bx + k, rr = by + tx;      // out is cols x rows
        if (cc < cols && rr < rows) out[(size_t)cc * rows + rr] = tile[tx][k];
    }
}

inline unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Partial sums of  sum(Q * W) - 2 sum(W * B^T) + trace(C)  in double, one per workgroup (fixed
// order inside a workgroup; the host adds the partials in order).  Q = W A (d x r), B is r x d.
__global__ __launch_bounds__(256) void surrogate_partial_kernel(int d, int r, const float* __restrict__ Q,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ B,
                                                                const float* __restrict__ Cagg,
                                                                double* __restrict__ partial) {
    __shared__ double red[256];
    const size_t n = (size_t)d * r;
    double acc = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / r), j = (int)(e % r);
        const double w = W[e];
        acc += w * ((double)Q[e] - 2.0 * (double)B[(size_t)j * d + i]);
    }
    if (Cagg != nullptr)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d; i += gridDim.x * 256) acc += (double)Cagg[(size_t)i * d + i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

}  // namespace

int launch_surrogate_partials(int d, int r, const float* Q, const float* W, const float* B, const float* Cagg,
                              double* partial, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(surrogate_partial_kernel, dim3(blocks), dim3(256), 0, s, d, r, Q, W, B, Cagg, partial);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_fill_uniform(float* out, size_t n, uint64_t seed, hipStream_t s) {
    if (n == 0) return ONMF_OK;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, out, n, seed);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_gather_columns(int d, int n, const float* X, int ldx, const long long* idx, int b,
                          float* Xb, hipStream_t s) {
    hipLaunchKernelGGL(gather_columns_kernel, dim3(grid_for((size_t)d * b)), dim3(256), 0, s, d, n, X,
                       ldx, idx, b, Xb);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_scatter_add_last(onmf_ctx* c, int r, int n, float* code, const long long* idx, int b,
                            const float* H, hipStream_t s) {
    ONMF_TRY(onmf_reserve(c, c->idx, (size_t)n * sizeof(int), "scatter scratch"));
    int* last = (int*)c->idx.p;
    hipLaunchKernelGGL(last_pos_init_kernel, dim3(grid_for(n)), dim3(256), 0, s, last, n);
    hipLaunchKernelGGL(last_pos_kernel, dim3(grid_for(b)), dim3(256), 0, s, last, n, idx, b);
    hipLaunchKernelGGL(scatter_last_kernel, dim3(grid_for((size_t)r * b)), dim3(256), 0, s, r, n, code,
                       last, idx, b, H);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_axpy(float* y, const float* x, float a, size_t n, hipStream_t s) {
    if (n == 0) return ONMF_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, a, n);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_axpby(float* out, const float* old, float keep, const float* part, float w, size_t n, hipStream_t s) {
    if (n == 0) return ONMF_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, old, keep, part, w, n);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_sub(float* out, const float* a, const float* b, size_t n, hipStream_t s) {
    if (n == 0) return ONMF_OK;
    hipLaunchKernelGGL(sub_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, a, b, n);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_sub_stack(float* out, const float* a, const float* b, size_t n, hipStream_t s) {
    if (n == 0) return ONMF_OK;
    hipLaunchKernelGGL(sub_stack_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, a, b, n);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_unfold3(int n0, int n1, int n2, const float* T, int mode, int transpose, float* out,
                   hipStream_t s) {
    hipLaunchKernelGGL(unfold3_kernel, dim3(grid_for((size_t)n0 * n1 * n2)), dim3(256), 0, s, n0, n1, n2,
                       T, mode, transpose, out);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_transpose(int rows, int cols, const float* in, float* out, hipStream_t s) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, rows, cols, in, out);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}
