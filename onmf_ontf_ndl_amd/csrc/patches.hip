// Image <-> patch-matrix kernels around the coder (HBM-bound gathers, no arithmetic to speak of).
//
// A patch is a k x k window of an h x w x ch image (C order), flattened as the reference does
// (image_reconstruction.py:187-188, :377-378: window.reshape(-1)): feature f = (x0 * k + x1) * ch + cc.
// The patch matrix is X (d x b, d = k*k*ch), one patch per column, patch index fastest in memory -
// so threads run along patches and every store / load of X is coalesced.
#include "common.h"

namespace {

__device__ __forceinline__ void patch_corner(const int* __restrict__ corners, int p, int grid_cols, int stride,
                                             int h, int w, int k, int& a, int& bb) {
    if (corners != nullptr) {
        a = corners[2 * p];
        bb = corners[2 * p + 1];
        // (a corner outside the image is the caller's error; clamped so that it cannot fault)
        a = min(max(a, 0), h - k);
        bb = min(max(bb, 0), w - k);
    } else {
        a = (p / grid_cols) * stride;
        bb = (p % grid_cols) * stride;
    }
}

// X[f][p] = img[a_p + x0][b_p + x1][cc]:  grid (ceil(b/256), d), one feature row per blockIdx.y
__global__ __launch_bounds__(256) void patch_extract_kernel(const float* __restrict__ img, int h, int w, int ch,
                                                            int k, const int* __restrict__ corners, int b,
                                                            int grid_cols, int stride, float* __restrict__ X,
                                                            int ldx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= b) return;
    const int f = blockIdx.y;
    const int cc = f % ch, x1 = (f / ch) % k, x0 = f / (ch * k);
    int a, bb;
    patch_corner(corners, p, grid_cols, stride, h, w, k, a, bb);
    X[(size_t)f * ldx + p] = img[((size_t)(a + x0) * w + (bb + x1)) * ch + cc];
}

// out[y][x][cc] = mean over the grid patches that cover pixel (y, x) of their value there; count =
// how many do.  One thread per output element, x fastest: for a fixed window offset (dy, dx)
// neighbouring threads read neighbouring patches of one feature row of R.
__global__ __launch_bounds__(256) void patch_fold_kernel(const float* __restrict__ R, int ldr, int h, int w, int ch,
                                                         int k, int grid_rows, int grid_cols, int stride,
                                                         float* __restrict__ out, float* __restrict__ count) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)h * w * ch;
    if (e >= n) return;
    const int x = (int)(e % w), y = (int)((e / w) % h), cc = (int)(e / ((size_t)w * h));
    float sum = 0.f;
    int cnt = 0;
    // patches in the reference's painting order (rows of the grid, then columns): dy, dx descending
    for (int dy = min(k - 1, y); dy >= 0; --dy) {
        const int ty = y - dy;
        if (ty % stride != 0) continue;
        const int i = ty / stride;
        if (i >= grid_rows) continue;
        for (int dx = min(k - 1, x); dx >= 0; --dx) {
            const int tx = x - dx;
            if (tx % stride != 0) continue;
            const int j = tx / stride;
            if (j >= grid_cols) continue;
            sum += R[(size_t)((dy * k + dx) * ch + cc) * ldr + (size_t)i * grid_cols + j];
            ++cnt;
        }
    }
    out[((size_t)y * w + x) * ch + cc] = cnt > 0 ? sum / (float)cnt : 0.f;
    if (count != nullptr && cc == 0) count[(size_t)y * w + x] = (float)cnt;
}

}  // namespace

int launch_patch_extract(const float* img, int h, int w, int ch, int k, const int* corners, int b, int grid_cols,
                         int stride, float* X, int ldx, hipStream_t s) {
    const int d = k * k * ch;
    hipLaunchKernelGGL(patch_extract_kernel, dim3((b + 255) / 256, d), dim3(256), 0, s, img, h, w, ch, k, corners, b,
                       grid_cols, stride, X, ldx);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}

int launch_patch_fold(const float* R, int ldr, int h, int w, int ch, int k, int grid_rows, int grid_cols, int stride,
                      float* out, float* count, hipStream_t s) {
    const size_t n = (size_t)h * w * ch;
    hipLaunchKernelGGL(patch_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, R, ldr, h, w, ch, k,
                       grid_rows, grid_cols, stride, out, count);
    return hipGetLastError() == hipSuccess ? ONMF_OK : ONMF_ERR_HIP;
}
