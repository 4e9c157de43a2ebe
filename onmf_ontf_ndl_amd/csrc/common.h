// Shared declarations for the gfx950 ONMF library (internal; the public C-ABI is include/onmf_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/onmf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// context: owns scratch only (slabs for split-K partials, the W^T X panel, per-sweep statistics)
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct onmf_ctx {
    int device = 0;
    std::string err;
    DevBuf slab;        // split-K partial products
    DevBuf cm;          // Cm = W^T X  (r x b)
    DevBuf gram;        // Gm = W^T W  (r x r)
    DevBuf hbuf[2];     // ping-pong code buffers (r x b)
    DevBuf stats;       // per-workgroup sweep statistics (double pairs)
    DevBuf small;       // r x r scratch x 2 (stop-test Grams) + dictionary-pass scratch
    DevBuf qt;          // dictionary pass: Qt = A^T W^T  (r x d)
    DevBuf sync;        // dictionary pass: cross-workgroup slots + counters
    DevBuf idx;         // scatter scratch
    std::vector<double> hstats;
    std::vector<float> hgram;
    void* pinned = nullptr;   // small pinned host staging
    size_t pinned_bytes = 0;
    // instrumentation: device time of the last dominant-kernel launches (filled on request)
    int profile = 0;
    int exact_stop = 0;
    int sweep_kernel = 0;
    std::vector<hipEvent_t> ev;     // pairs of events bracketing sweep launches
    int ev_used = 0;
    DevBuf dbuf;                    // H - H_prev for the exact stopping test (r x b)
    DevBuf screen;                  // stopping-test screen state (ScreenLayout)
    DevBuf slab_side;               // split-K slabs of products issued on the side stream
    hipStream_t side = nullptr;     // dictionary pass runs here, beside the code solve
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int overlap_dict = 1;
    int screen_stop = 1;
    int screen_r = -1;              // r of the probes currently held in `screen`
    DevBuf ddelta;                  // dictionary pass: atom deltas (d x r), one-workgroup kernel
    int dict_kernel = 0;            // 0 = by shape, 1 = cooperative workgroups, 2 = one workgroup
    bool dict_coop_last = false;    // the last dictionary pass used the cooperative kernel (its timeout flag is live)
};

int onmf_fail(onmf_ctx* c, int code, const char* fmt, ...);
int onmf_reserve(onmf_ctx* c, DevBuf& b, size_t bytes, const char* what);
int onmf_reserve_pinned(onmf_ctx* c, size_t bytes);

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return onmf_fail((ctx), ONMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

#define ONMF_TRY(expr)                                                                          \
    do {                                                                                        \
        int _rc = (expr);                                                                       \
        if (_rc != ONMF_OK) return _rc;                                                         \
    } while (0)

// ---------------------------------------------------------------------------------------------
// kernels' host launchers (defined in the .hip files)
// ---------------------------------------------------------------------------------------------
// C = beta*Cin + alpha * op(A) op(B).  a_km: A(m,k)=A[k*lda+m] else A[m*lda+k];
// b_kn: B(k,n)=B[k*ldb+n] else B[n*ldb+k].  Cin may be null when beta == 0; Cin may alias C.
int launch_gemm(onmf_ctx* c, bool a_km, bool b_kn, int M, int N, int K, const float* A, int lda,
                const float* B, int ldb, float alpha, float beta, const float* Cin, float* C,
                int ldc, hipStream_t s, DevBuf* slab = nullptr);

// Device pointers of the stopping-test screen (screen.hip); all inside one allocation.
struct ScreenDev {
    const float* ublk;   // NB swizzled 16x16 blocks of U^T (16 unit probe vectors, r_pad long)
    const float* v;      // r_pad: positive vector of the Collatz-Wielandt bound
    float* zacc;         // ZREPL x r_pad x 16: power-step accumulator
    float* p2acc;        // REPL x 16: probe energies
    float* yacc;         // REPL x 256: H (H^T v)
    double* sums;        // REPL x {sum (H_new - H_old)^2, sum H_old^2}
    const int* flag;     // 0 = keep sweeping
};

// One Gauss-Seidel sweep over all rows: Hout = sweep(Hin).  stats (device, 2 doubles per
// workgroup) receives sum((Hout-Hin)^2) and sum(Hin^2) partials; *n_groups = number of pairs.
int launch_sweep_tile_ct1(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                          float* Hout, float alpha, float damp, int exact_cd, double* stats,
                          int* n_groups, hipStream_t s, const ScreenDev* screen);
int launch_sweep_tile_ct2(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                          float* Hout, float alpha, float damp, int exact_cd, double* stats,
                          int* n_groups, hipStream_t s, const ScreenDev* screen);
int launch_sweep(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                 float* Hout, float alpha, float damp, int exact_cd, double* stats, int* n_groups,
                 hipStream_t s, const ScreenDev* screen = nullptr);
bool sweep_tile_covers(int r, int b);

// Stopping-test screen: state layout, initialisation and the per-sweep decision kernel.
constexpr int SCREEN_MAX_SWEEPS = 64;
constexpr int SCREEN_REPL = 8;       // replicas of the sweep kernel's accumulators (atomic contention)
constexpr int SCREEN_ZREPL = 4;      // replicas of the power-step accumulator
struct ScreenLayout {           // offsets in floats into ctx->screen
    static constexpr int UBLK = 0;                                 // 16 * 256
    static constexpr int V = 4096;                                 // 256
    static constexpr int ZACC = V + 256;                           // ZREPL x 256 x 16
    static constexpr int P2ACC = ZACC + SCREEN_ZREPL * 4096;       // REPL x 16
    static constexpr int YACC = P2ACC + SCREEN_REPL * 16;          // REPL x 256
    static constexpr int SUMS = YACC + SCREEN_REPL * 256;          // REPL x 2 doubles (8-byte aligned)
    static constexpr int FLAG = SUMS + SCREEN_REPL * 4;            // int
    static constexpr int UBPREV = FLAG + 1;        // float: upper bound of sigma_max(H before the sweep)
    static constexpr int SKIP = FLAG + 2;          // int: the next sweep's probe power step is skipped
    static constexpr int AVEC = FLAG + 4;          // 16: last Ritz coefficient vector (warm start)
    static constexpr int REC = AVEC + 16;          // SCREEN_MAX_SWEEPS x 4 floats
    static constexpr int TOTAL = REC + 4 * SCREEN_MAX_SWEEPS;
};
static_assert(ScreenLayout::SUMS % 2 == 0, "double alignment");
ScreenDev screen_pointers(float* base);
int launch_screen_init(onmf_ctx* c, float* base, int r, hipStream_t s);
int launch_screen_decide(onmf_ctx* c, float* base, int r, int sweep_index, float tol, float rank_cap,
                         hipStream_t s);
int launch_screen_clear_flag(onmf_ctx* c, float* base, hipStream_t s);
int launch_screen_refresh(onmf_ctx* c, float* base, int r, int b, const float* Hnew, const float* Hold,
                          hipStream_t s);

int launch_dict_pass(onmf_ctx* c, int d, int r, const float* Win, const float* A, const float* B,
                     float* Wout, hipStream_t s);

int launch_fill_uniform(float* out, size_t n, uint64_t seed, hipStream_t s);
int launch_gather_columns(int d, int n, const float* X, int ldx, const long long* idx, int b,
                          float* Xb, hipStream_t s);
int launch_scatter_add_last(onmf_ctx* c, int r, int n, float* code, const long long* idx, int b,
                            const float* H, hipStream_t s);
int launch_axpy(float* y, const float* x, float a, size_t n, hipStream_t s);
int launch_sub(float* out, const float* a, const float* b, size_t n, hipStream_t s);
int launch_unfold3(int n0, int n1, int n2, const float* T, int mode, int transpose, float* out,
                   hipStream_t s);
int launch_transpose(int rows, int cols, const float* in, float* out, hipStream_t s);

// host: largest eigenvalue of a symmetric PSD n x n matrix (row-major float), in double
double host_lambda_max(const float* S, int n);
